#!/bin/bash
# the whole GPU suite + smoke + the default bench line
out=gpurun_out/r06_full; rm -rf $out; mkdir -p $out
timeout -k 10 1150 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -12 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; j=json.load(open('$out/bench_default.json')); print(j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j['roofline']['frac'], j['roofline'].get('frac_by_counters'), j['accounting_violations'])"
