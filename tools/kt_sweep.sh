for v in "" k10 k8 k6; do
  if [ -n "$v" ]; then export JPEG_AMD_LIB=$PWD/build_var/libjpeg_amd_$v.so; fi
  echo "== variant ${v:-base}"
  for c in metric_4k_420_q90:128 c3_8k_gray12_q95:16 c4_8k_420_q95:16; do timeout -k 10 200 python tools/ktimes.py ${c%%:*} ${c##*:} | tail -2 | cut -c1-420 || exit 1; done
  KT_RI=480 timeout -k 10 200 python tools/ktimes.py c2_4k_444_q90 32 | tail -2 | cut -c1-420 || exit 1
done
