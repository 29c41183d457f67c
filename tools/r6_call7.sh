cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c7; mkdir -p $out
timeout 600 python -m pytest tests/test_graph_module_gpu.py -x -q -k "autograd_step or accumulation" > $out/pytest.txt 2>&1; tail -15 $out/pytest.txt
timeout 600 python bench.py --use-beta > $out/bench_beta.json 2> $out/bench_beta.err; tail -c 1500 $out/bench_beta.err; cat $out/bench_beta.json | cut -c1-3000
