#!/bin/bash
# quick GPU check used during development: the GPU test-suite, then the default bench without the CPU leg
set -e
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -5
timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
