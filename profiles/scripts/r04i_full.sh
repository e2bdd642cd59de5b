#!/bin/bash
# round 4, session 3 baseline: the whole -m gpu suite, smoke, the default bench line (with training extras) and its kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -q -x -m gpu > $O/r04i_full_pytest.txt 2>&1; tail -5 $O/r04i_full_pytest.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r04i_bench_default.json; cut -c1-700 $O/r04i_bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r04i_kt -o t -- python $R/bench.py --no-cpu-baseline > $O/r04i_kt.log 2>&1
python $R/profiles/summarize_rocprof.py $O/r04i_kt/t_results.db > $O/r04i_default_kernel_stats.txt; rm -rf $O/r04i_kt
head -40 $O/r04i_default_kernel_stats.txt
