#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout -k 10 1100 python tools/fuzz_tick_vs_oracle.py 4 24 300 > gpurun_out/fuzz/fuzz_tick.jsonl 2> gpurun_out/fuzz/fuzz_tick.err || { tail -8 gpurun_out/fuzz/fuzz_tick.err; tail -3 gpurun_out/fuzz/fuzz_tick.jsonl; exit 1; }
cat gpurun_out/fuzz/fuzz_tick.jsonl | cut -c1-400
