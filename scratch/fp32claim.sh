python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "non_finite" 2>&1 | tail -3
python -m pytest tests/test_graph_gpu.py -x -q -m gpu -k "channel_spread" -s 2>&1 | grep -E "^fp32|^bf16x3|^f16x2|passed|failed|Error|assert" | cut -c1-700
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['roofline']['achieved'])"
done
