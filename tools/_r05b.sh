mkdir -p gpurun_out/r05b
timeout 600 python tools/probe/side_stream_determinism.py 8 > gpurun_out/r05b/side_stream.txt 2>&1; tail -30 gpurun_out/r05b/side_stream.txt | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r05b/bench.json').readline()); print(d['ms_per_step'], json.dumps(d['cpu_baseline'])[:1200])"
