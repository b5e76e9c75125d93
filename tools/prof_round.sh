#!/bin/bash
# The round's evidence in one GPU call (run through gpurun from the repo root):   tools/prof_round.sh r05
#   1. rocprofv3 --kernel-trace --stats of the driver's own command (python3 bench.py --steps 20 --warmup 5)
#   2. the same command without the profiler, with --detail
#   3. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no other trace domain beside --kernel-trace)
#   4. the 1-rank shard rehearsal (tools/shard_rehearsal.py)
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -e
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/default -o $TAG --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err && echo default-done
python3 $R/tools/profile_digest.py by-grid $(find $O/default -name '*kernel_trace.csv' | head -1) > $O/kernel_stats_by_grid.csv
cp $(find $O/default -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench_default.csv
find $O -name '*kernel_trace.csv' -delete
python3 $R/bench.py --steps 20 --warmup 5 --detail $O/bench_plain_detail.json > $O/bench_plain.json 2> $O/bench_plain.err && echo plain-done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o $TAG --output-format csv -- python3 $R/bench.py --no-cpu --no-extra --no-host --steps 20 --warmup 5 > $O/pmc_fetch.json 2> $O/pmc_fetch.err && echo fetch-done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o $TAG --output-format csv -- python3 $R/bench.py --no-cpu --no-extra --no-host --steps 20 --warmup 5 > $O/pmc_write.json 2> $O/pmc_write.err && echo write-done
python3 $R/tools/pmc_traffic_summary.py $O/fetch $O/write > $O/pmc_traffic.json || echo pmc-summary-failed
find $O -name '*kernel_trace.csv' -delete
find $O -name '*counter_collection.csv' -size +20M -delete
python3 $R/tools/shard_rehearsal.py $O/shard_rehearsal.json > $O/shard.log 2>&1 && echo shard-done
ls -la $O
