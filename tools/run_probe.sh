python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "2048 20 W 1" "2048 20 W 16"; do echo "== $cfg"; python tools/map_probe.py $cfg 2>&1 | grep -v "^W2026" | tail -1; done
bash tools/prof_map.sh r01e 1024 20 W 2>&1 | grep -v "^W2026" | tail -12
