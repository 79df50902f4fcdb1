#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout -k 10 1100 python tools/fuzz_vs_oracle.py 40 256 > gpurun_out/fuzz/fuzz.jsonl 2> gpurun_out/fuzz/fuzz.err || { tail -5 gpurun_out/fuzz/fuzz.err; tail -3 gpurun_out/fuzz/fuzz.jsonl; exit 1; }
tail -1 gpurun_out/fuzz/fuzz.jsonl
