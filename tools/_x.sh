#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/_prof_host.py 2>&1 | tail -45
