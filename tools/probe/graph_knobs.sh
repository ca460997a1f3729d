mkdir -p gpurun_out/g11
run() { echo "== $*" >> gpurun_out/g11/knobs.log; env "$@" timeout 300 python bench.py --steps 12 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],2), d['config']['step_graph'])" >> gpurun_out/g11/knobs.log 2>&1; }
run X=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run ET_WGRAD_STREAM=0
cat gpurun_out/g11/knobs.log
