#!/bin/bash
# Round 6, call 26: api.load() imports torch before the library (call 25: build() + smoke() in ONE process loaded the library first and ended up with two HIP
# runtimes).  smoke() both ways, then the whole GPU suite in one command on the final commit.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 250 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/r06_smoke_call26_build_then_smoke.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2 | tee $O/r06_smoke_call26.txt
timeout 700 python -m pytest tests -m gpu -x -q --durations=5 > $O/r06_gputests_call26.txt 2>&1
echo "pytest rc=$?" | tee -a $O/r06_gputests_call26.txt; tail -4 $O/r06_gputests_call26.txt
