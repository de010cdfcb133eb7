#!/bin/bash
# Work-distribution A/B on one box: build the variants first, e.g.
#   tools/build_variants.sh "p0:-DQBP_WORK_CHUNK_FIXED=1" "p4:"
#   gpurun -- 'bash tools/ab_chunk.sh'      -> profiles/r02_ab_work_chunk.txt
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 800 python tools/ab_run.py tools/ab_work.py > gpurun_out/ab_work.txt 2>&1
cat gpurun_out/ab_work.txt
