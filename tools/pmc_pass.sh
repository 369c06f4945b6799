#!/bin/bash
# usage: tools/pmc_pass.sh <name> <planes> <counters...>   (one rocprofv3 --pmc pass over tools/pmc_chain.py)
name=$1; planes=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcx_$name -o $name -- python3 $GRAFT_REPO_ROOT/tools/pmc_chain.py $planes 2 > $GRAFT_REPO_ROOT/gpurun_out/pmcx_$name.log 2>&1
