export LDC_LIB_PATH=ladcast_amd/libladcast_hip_ab.so
for rep in 1 2; do
for G in "" 216 240 252 256 192; do
  if [ -z "$G" ]; then unset LDC_BF16X3_G; else export LDC_BF16X3_G=$G; fi
  python3 tools/gemm_g_sweep.py 2250 1536 7680 2>&1 | grep -v amdgpu.ids
done; done
unset LDC_BF16X3_G
echo "--- dual out (1800+450) x 1536 x 1536, G auto / 240 / 256"
for G in "" 240 256; do
  if [ -z "$G" ]; then unset LDC_BF16X3_G; else export LDC_BF16X3_G=$G; fi
  python3 tools/gemm_g_sweep.py 1800 1536 1536 450 1536 1536 2>&1 | grep -v amdgpu.ids
done
