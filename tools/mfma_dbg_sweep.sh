export SAFE_HIP_MFMA_PREF=0 SAFE_HIP_MFMA_DBG_NOMEMBERS=1
for d in 0 1 3 8 4 11 15; do echo -n "dbg=$d: "; SAFE_HIP_MFMA_DBG=$d timeout 250 python tools/bench_big.py quant 2048 256 2>/dev/null | tail -1 | cut -c1-70; done
unset SAFE_HIP_MFMA_PREF SAFE_HIP_MFMA_DBG_NOMEMBERS
for d in 0 1 3 8; do echo -n "real, dbg=$d: "; SAFE_HIP_MFMA_DBG=$d timeout 250 python tools/bench_big.py quant 2048 256 2>/dev/null | tail -1 | cut -c1-70; done
