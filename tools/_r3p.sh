set -o pipefail
cd /root/repo
bash tools/gpu_profile.sh headline "8 96 4096 128" > gpurun_out/prof_headline.log 2>&1 || { tail -20 gpurun_out/prof_headline.log; exit 1; }
bash tools/gpu_profile.sh gpt3_1k "4 96 1024 128" > gpurun_out/prof_gpt3_1k.log 2>&1 || { tail -20 gpurun_out/prof_gpt3_1k.log; exit 1; }
bash tools/gpu_profile.sh d64 "16 32 2048 64" > gpurun_out/prof_d64.log 2>&1 || { tail -20 gpurun_out/prof_d64.log; exit 1; }
grep -h '"metric"' gpurun_out/prof/*/bench_prof.json | cut -c1-400
