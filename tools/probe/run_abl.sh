mkdir -p gpurun_out/g2
SH="256,256,3,1,40,64;1024,1024,1,1,20,64;512,512,3,1,20,64"
ET_CONV_PP=0 python tools/probe/abl_conv.py "$SH" 2>&1 | grep ABL >> gpurun_out/g2/abl.log
python tools/probe/abl_conv.py "$SH" 2>&1 | grep ABL >> gpurun_out/g2/abl.log
for n in 20 21 22 23 24 25 26 27 28; do ET_HIP_LIB=tools/probe/libet_abl$n.so timeout 120 python tools/probe/abl_conv.py "$SH" 2>&1 | grep ABL >> gpurun_out/g2/abl.log; done
ET_HIP_LIB=tools/probe/libet_abl9.so python tools/probe/ts_conv.py 256 256 3 1 40 64 2>&1 | grep TS >> gpurun_out/g2/abl.log
ET_CONV_PP=0 ET_HIP_LIB=tools/probe/libet_abl9.so python tools/probe/ts_conv.py 256 256 3 1 40 64 2>&1 | grep TS >> gpurun_out/g2/abl.log
cat gpurun_out/g2/abl.log
