# GPU session r5t: store-stream lab — block geometry / position mapping of a 12-byte-tuple store stream (tools/store_lab.hip)
set -u
O=gpurun_out/r5t
mkdir -p $O
timeout 300 tools/store_lab > $O/store_lab.txt 2>&1; echo "rc=$?"; cat $O/store_lab.txt
