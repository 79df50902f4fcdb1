#!/bin/bash
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so timeout -k 10 200 python tools/stamps_tick.py 8192 200 kin | cut -c1-1200
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so timeout -k 10 200 python tools/stamps_tick.py 8192 200 kin 5.0 | cut -c1-1200
WCQP_LIB_PATH=$PWD/walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so timeout -k 10 200 python tools/stamps_tick.py 8192 150 kin | cut -c1-1200
