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
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_stats.json 2> $out/stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/write.err
python3 scripts/pmc_traffic.py $out/fetch $out/write gpurun_out/profiles_$tag/${tag}_pmc_traffic.json
cp $(ls $out/stats/*/*_kernel_stats.csv | head -1) gpurun_out/profiles_$tag/${tag}_bench_kernel_stats.csv
cp $out/bench_stats.json gpurun_out/profiles_$tag/${tag}_bench.json
python3 scripts/prof_summary.py $out/stats
rm -rf $out/fetch $out/write   # counter CSVs are tens of MB
