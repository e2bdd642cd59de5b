R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -q -x -m gpu > $O/r03_full_pytest.txt 2>&1; tail -5 $O/r03_full_pytest.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r03_full_bench.json; cut -c1-600 $O/r03_full_bench.json
