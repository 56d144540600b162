set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 70 python $R/bench.py --also-sup 0 --through-host 0 2>$O/bench.err | tail -1 > $O/r05_v_bench_hac_only.json; cut -c1-600 $O/r05_v_bench_hac_only.json
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --model hac --steps 3 --warmup 1 --also-sup 0 --through-host 0 --no-cpu-baseline --profile-run > $O/stats.log 2>&1
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r05_v_kernel_stats_hac_n16384.csv && head -9 $O/r05_v_kernel_stats_hac_n16384.csv | cut -c1-150
rm -rf $O/stats
