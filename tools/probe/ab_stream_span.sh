for span in "" 64 100 128 200; do for d in 0 256; do
  echo -n "span=$span dbg=$d: "; SAFE_HIP_BITS_SPAN=$span SAFE_HIP_BITS_DBG=$d timeout 120 python tools/bits_ablate.py --one 1000 2>/dev/null | tail -1
done; done
