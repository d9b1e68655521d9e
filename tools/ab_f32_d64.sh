# fp32 tensors at d = 64: 128-rows-per-wave layout (FLASH_HIP_TALL=1, default) against 64 rows per wave, in the
# kernel and through the conversion pass (selftest benchf32 prints both entries)
for round in 1 2; do
for t in 0 1; do
  for s in "16 32 512 64 300" "16 32 1024 64 150" "16 32 2048 64 60" "16 32 4096 64 20"; do
    FLASH_HIP_TALL=$t timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep "flash_fwd_f32" | sed "s|^|[tall=$t] |" || exit 1
  done
done
done
