"""bench.py --gpus N started WITHOUT a launcher must start its N ranks itself (the driver's contract; the fan-out the
reference does with its worker map, src/DomainDecomposition/DDParallel.jl:87-105,133-139)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_launcher_refuses_more_ranks_than_gpus_unless_sharing_is_allowed():
    """No GPU initialised, no rank started: the parent only counts devices."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       env=_clean_env(HIP_VISIBLE_DEVICES="0"), timeout=120)
    assert r.returncode != 0
    assert "only 1 GPU(s) visible" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["ghost", "halo"])
def test_bench_gpus2_from_a_bare_python_call(form):
    """Two fresh rank processes sharing the box's one GPU (host-staged plug-in transport), started by bench.py itself;
    rank 0's single JSON line comes back with n_gpus = 2 and the launcher's N = 1 reference - in the ghost-layer form (default:
    mg_ghost_*) and in the halo form of rounds 2-4 (mg_dist_*)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "32", "--steps", "3",
                        "--warmup", "1", "--sharded-form", form], capture_output=True, text=True, env=_clean_env(MG_BENCH_ALLOW_SHARE="1"),
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["launcher"]["ranks_started"] == 2 and out["launcher"]["shared_gpu"] is True
    assert out["n1_reference"]["value"] > 0 and out["parallel_efficiency_vs_n1"] is not None
    assert ("ghost layers" if form == "ghost" else "native C++ sequencer") in out["config"]["parallelism"]
    if form == "ghost":
        assert out["ghost"]["exchanges_per_step"] > 0 and out["transport"].startswith("plug-in")
    assert 0 < out["relres_after_steps"] < 1e-2
