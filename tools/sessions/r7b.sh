# GPU session r7b: where a small host-staged delivery pass (what GpuShared's batcher issues at config 2) spends its 2.2 ms
set -u
O=$PWD/gpurun_out/r7b
mkdir -p $O
timeout 900 python3 tools/deliver_pass_profile.py 2600 300 > $O/profile_2600.txt 2> $O/profile_2600.err; echo "rc=$?"; cat $O/profile_2600.txt; tail -3 $O/profile_2600.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/deliver_pass_profile.py 2600 100 > $O/profile_under_rocprofv3.txt 2> $O/profile_under_rocprofv3.err; echo "prof rc=$?"
for k in kernel_stats hip_api_stats; do f=$(find $O/trace -name "*${k}.csv" | head -1); [ -n "$f" ] && cp "$f" $O/deliver_pass_${k}.csv && head -14 "$f" | cut -c1-60,200-330; done; rm -rf $O/trace
