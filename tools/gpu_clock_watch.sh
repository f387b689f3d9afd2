#!/bin/bash
# Samples sclk / socket power with rocm-smi while bench.py runs a long C2 MD run (GPU box only): is the fp32 MFMA load
# power-capped below the 2.4 GHz boost clock the 157.3 TF peak assumes?
python bench.py --no-cpu-baseline --no-secondary --steps 6000 --warmup 20 > /tmp/b.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed -e 's/.*level: //' -e 's/.*(W): /W=/' | tr '\n' ' '; echo
  sleep 1
done | awk '{print NR": "$0}' | grep -v "(1[0-9][0-9]Mhz)\|([0-9][0-9]Mhz)" | tail -25
python -c "import json; d=json.load(open('/tmp/b.json')); print(d['ms_per_step'], d['roofline']['frac'], d['config']['edges_per_step'])"
