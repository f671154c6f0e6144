mkdir -p gpurun_out
(python scripts/diag_pcg.py 2>&1 | grep "GMRES"; MG_NO_MGS_CHAIN=1 python scripts/diag_pcg.py 2>&1 | grep "GMRES") | tee gpurun_out/diag_fgmres_ab.txt
