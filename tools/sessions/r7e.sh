# GPU session r7e: the HIP calls and kernels of a small host-staged delivery pass (2 600 publishes of config 2 per call)
set -u
O=$PWD/gpurun_out/r7e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/deliver_pass_profile.py 2600 200 > $O/profile_under_rocprofv3.txt 2> $O/profile_under_rocprofv3.err; echo "prof rc=$?"
cat $O/profile_under_rocprofv3.txt | cut -c1-400
for k in kernel_stats hip_api_stats; do f=$(find $O/trace -name "*${k}.csv" | head -1); [ -n "$f" ] && cp "$f" $O/deliver_pass_${k}.csv && head -24 "$f" | cut -c1-110; done; rm -rf $O/trace
