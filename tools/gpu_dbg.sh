#!/bin/bash
OUT=gpurun_out/dbg; mkdir -p $OUT
python -m pytest tests/test_plugin_minihost_gpu.py "tests/test_multirank_gpu.py::test_two_ranks_shard_one_recording_on_the_engines[metop_ahrpt-252]" -m gpu -q --tb=short 2>&1 | grep -v "^\[sdhip\]" | cut -c 1-600 > $OUT/pytest.txt
tail -120 $OUT/pytest.txt
