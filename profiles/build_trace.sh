#!/bin/bash
# profiling build of the engine with in-kernel phase counters: build_variants/libmlst_trace.so (use with MLST_LIB=...)
set -e
cd /root/repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMLST_EXT_TRACE=${MLST_TRACE_LEVEL:-1} -Wno-unused-result -Wno-unused-value -Wno-parentheses -Wno-pass-failed -Iinclude -o build_variants/libmlst_trace.so metamlst_amd/csrc/mlst_engine.hip
