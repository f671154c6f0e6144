"""multigrid.jl_amd - MI355X-native multigrid cycle behind the MGparam / MGsetup / solveMG surface
of JuliaInv/Multigrid.jl.  Host side (this package): hierarchy setup on the CPU, as in the reference.
Device side (csrc/): the cycle itself, hand-written HIP for gfx950, reached through the C ABI in
include/mgvcycle.h.  Import as ``import multigrid_jl_amd`` (shim at the repo root: the directory
name carries a dot).
"""
from .mgdef import (MGparam, getMGparam, hierarchyExists, destroyCoarsestLU, copySolver, clear_)
from .mgsetup import (MGsetup, getRelaxPrec, getSPAIprec, adjustMemoryForNumRHS, replaceMatrixInHierarchy,
                      transposeHierarchy, defineCoarsestAinv, multilevelOperatorConstructor,
                      getMultilevelOperatorConstructor, galerkin)
from .transfer_operators import (getFWInterp, get1DFWInterp, restrictCellCenteredVariables, restrictNodalVariables,
                                 getRestrictionCellCentered)
from .sa_amg import (SA_AMGsetup, getAggregation, getStrengthMatrix, neighborhoodAggregationNew, aggrArray2P)
from .solve_funcs import solveMG, solveCG_MG, solveBiCGSTAB_MG, solveGMRES_MG, recursiveCycle, SpMatMul, getMultigridPreconditioner, to_device
from .operators import (getRegularMesh, getNodalGradientMatrix, getNodalLaplacianMatrix,
                        getNodalDivSigGradMatrix, poisson_shifted, anisotropic_divsiggrad, seeded_rhs)
from .wrappers import (MGsolver, getMGsolver, getSA_AMGsolver, solveLinearSystem_, setupSolver, copySolverWrapper,
                       clearSolver_)
from .par_relax import (hybridKaczmarz, getHybridKaczmarz, setupHybridKaczmarz, getHybridKaczmarzPrecond,
                        applyHybridKaczmarz)
from .dd_indices import (getIndicesOfCellsArray, getNodalIndicesOfCell, getOriginalBoundingBoxCells, getBoxWithOverlap,
                         cs2loc, loc2cs)
from . import device

__all__ = [n for n in dir() if not n.startswith("_")]

# src/ParallelJuliaSolver is a sub-module of the reference package too (Multigrid.ParallelJuliaSolver)
from . import parallel_julia_solver as ParallelJuliaSolver
