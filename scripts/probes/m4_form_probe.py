import os, sys
sys.path.insert(0, ".")
os.environ.update(MG_ROWCLASS_MIN_ROWS="0", MG_NO_SMALL="1", MG_ROWCLASS_MAX_PASSES="64", MG_ROWCLASS_MIN_COVER="0.05", MG_MARCH_MIN_WG="0", MG_MARCH_MAX_LEN="64", MG_DEBUG_FORMAT="1")
import numpy as np
import multigrid_jl_amd as mg
for cells in ([24, 20, 16], [24, 19, 16]):
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, 3, 8, 3, 0.0, "Jac", 0.8, 2, 1, "W", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 4)
    h = mg.to_device(p)
    print(cells, "four-stage:", h.four_stage_form(1), "sweep-residual:", h.sweep_residual_form(1), flush=True)
    mg.clear_(p)
