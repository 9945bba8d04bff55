# GPU session r6g: is walk order still worth building on today's walk kernel? (tools/walk_order_lab.py through the product API: generator order vs
# topics sorted by string vs shuffled; config 2 at full size, config 3 at 3/10 scale)
set -u
O=$PWD/gpurun_out/r6g
mkdir -p $O
timeout 600 python3 tools/walk_order_lab.py 2 1.0 10 > $O/walk_order_lab_config2.txt 2>&1; cat $O/walk_order_lab_config2.txt
timeout 900 python3 tools/walk_order_lab.py 3 0.3 3 > $O/walk_order_lab_config3_scale0.3.txt 2>&1; cat $O/walk_order_lab_config3_scale0.3.txt
