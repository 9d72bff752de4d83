#!/bin/bash
# One round's judged profile set, on the GPU box: scripts/profile_round.sh <tag>   (e.g. r01k)
#   gpurun_out/<tag>/stats   rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3`
#   gpurun_out/<tag>/fetch, write   separate --pmc passes (never combined with other trace domains)
# and copies the summaries into profiles/ (the copy in gpurun_out/profiles_<tag>/ is what travels back).
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out gpurun_out/profiles_$tag
# a fresh box is still paging the image in: the first process runs host-bound (19-24 ms/step under the profiler
# measured twice), so one throw-away run comes first
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/prewarm.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_stats.json 2> $out/stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/write.err
python3 scripts/pmc_traffic.py $out/fetch $out/write gpurun_out/profiles_$tag/${tag}_pmc_traffic.json
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) gpurun_out/profiles_$tag/${tag}_bench_kernel_stats.csv
cp $out/bench_stats.json gpurun_out/profiles_$tag/${tag}_bench.json
python3 scripts/prof_summary.py $out/stats
rm -rf $out/fetch $out/write   # counter CSVs are tens of MB
# un-profiled bench lines of every model family (the default one with the CPU baseline)
timeout -k 10 400 python3 bench.py > gpurun_out/profiles_$tag/${tag}_bench_unprofiled.json 2> $out/unprofiled.err
timeout -k 10 300 python3 bench.py --model attention_unet --no-cpu-baseline > gpurun_out/profiles_$tag/${tag}_bench_attention_unet.json 2> $out/att.err
timeout -k 10 300 python3 bench.py --model resnext_unet --size 512 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/profiles_$tag/${tag}_bench_resnext_unet.json 2> $out/res.err
timeout -k 10 300 python3 bench.py --model trans_unet --batch 32 --patch-size 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/profiles_$tag/${tag}_bench_trans_unet_p4.json 2> $out/tr4.err
timeout -k 10 300 python3 bench.py --model trans_unet --batch 32 --patch-size 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/profiles_$tag/${tag}_bench_trans_unet_p2.json 2> $out/tr2.err
# kernel statistics of the TransUNet step (BASELINE configs[4])
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_trans -- python3 bench.py --model trans_unet --batch 32 --patch-size 4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/stats_trans.err
cp $(ls $out/stats_trans/*/*_kernel_stats.csv | head -1) gpurun_out/profiles_$tag/${tag}_bench_trans_unet_p4_kernel_stats.csv
rm -rf $out/stats_trans
# ... and of the residual U-Net step (BASELINE configs[3])
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_res -- python3 bench.py --model resnext_unet --size 512 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/stats_res.err
cp $(ls $out/stats_res/*/*_kernel_stats.csv | head -1) gpurun_out/profiles_$tag/${tag}_bench_resnext_unet_kernel_stats.csv
rm -rf $out/stats_res
