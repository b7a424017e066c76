for n in 8 12; do
  echo "== MPX_LOW_MAX_CHUNKS=$n"
  MPX_LOW_MAX_CHUNKS=$n CASE=3 timeout 600 python tools/r4_light_ab.py "-DMPX_LOW_MAX_CHUNKS=$n" 2>&1 | grep -v amdgpu.ids | tail -4
done
