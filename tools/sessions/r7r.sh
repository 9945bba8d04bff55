# GPU session r7r: what a delivery pass of 3 600 publishes costs when 1 .. 4 host threads issue passes at once (the batcher's passes in flight)
set -u
O=$PWD/gpurun_out/r7r
mkdir -p $O
timeout 900 python3 tools/deliver_pass_profile.py 3600 600 4 > $O/profile_threads.txt 2> $O/profile_threads.err; echo "rc=$?"; tail -5 $O/profile_threads.txt | cut -c1-250
