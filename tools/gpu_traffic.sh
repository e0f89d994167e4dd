#!/bin/bash
# PMC traffic + SQ counter passes of the bench workload -> gpurun_out/$1/{traffic.csv,sq.txt}
OUT=gpurun_out/${1:-t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_traffic.py run > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_traffic.py run > /dev/null 2>&1
python3 tools/pmc_traffic.py report $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.csv 2>&1
cat $OUT/traffic.csv
