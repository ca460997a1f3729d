#!/bin/bash
# does keeping the pieces' address VGPRs untouched until after the step's MFMAs remove the buffer LDS-DMA mismatches?
mkdir -p gpurun_out
ET_PROBE_LIBS=default,tools/probe/libet_nolgkm.so,default,tools/probe/libet_nolgkm.so,default,tools/probe/libet_nolgkm.so timeout 900 python tools/probe/buf_nops.py 60 > gpurun_out/r06_keep.txt 2>&1
cat gpurun_out/r06_keep.txt
