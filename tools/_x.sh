#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
HYP_TRIAL_DBG=1 timeout 600 python bench.py --config 3b 2> /tmp/err.txt | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('phases_ms_per_step'), d.get('search_trials_per_step'), d.get('per_step'))"
grep "^\[trial\]" /tmp/err.txt | sed 's/[0-9.e+-]\+/N/g' | sort | uniq -c | sort -rn | head
