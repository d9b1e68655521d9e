# usage: ab.sh dirA dirB ... ; benches each lib twice, interleaved, on this device
for rep in 1 2; do
for d in "$@"; do
  if [ "$d" = "cur" ]; then L=flash-attention_amd/lib; else L=gpurun_tmp/$d; fi
  LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH timeout -k 10 60 tests/native/selftest bench 4 96 4096 128 20 11 | grep bench | sed "s/^/$d: /"
done
done
