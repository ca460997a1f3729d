SH="256,256,3,1,40,64;512,512,3,1,20,64"
echo -n "default "; python tools/probe/abl_conv.py "$SH" 2>&1 | grep ABL | cut -c1-260
for n in 21 23 24 29; do echo -n "abl$n "; ET_HIP_LIB=tools/probe/libet_abl$n.so timeout 120 python tools/probe/abl_conv.py "$SH" 2>&1 | grep ABL | cut -c1-260; done
