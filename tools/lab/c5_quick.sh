#!/bin/bash
# Development: config 5 with the product library, then the stamps of one build variant (default srt)
timeout 300 python3 tools/perf_configs.py 5 2>&1 | grep "^config\|oracle"
FASP_HIP_LIB=lab_build/libfasp_hip_${1:-srt}.so timeout 300 python3 tools/perf_configs.py 5 2>&1 | grep "spcg_dpp" | sed -n "300,302p"
