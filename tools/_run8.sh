cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in 1024 2048 4096; do
echo "windows $w: $(timeout 600 python bench.py --windows $w --steps 40 --warmup 5 --cpu-windows 0 --no-cold-start --no-mode-a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_step']*1024/$w,3), 'norm', round(d['value_with_normalisation']), d['nan_outputs'])")"
done
python tools/bench_stream.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['run0'], d['run1'], d['warm_same_stream'])"
python tests/fuzz_sweep.py 6490 6498 2>&1 | grep -v amdgpu | tail -3; python tests/fuzz_sweep.py 20450 20460 2>&1 | grep -v amdgpu | tail -3
