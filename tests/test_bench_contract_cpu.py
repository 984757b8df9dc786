

def test_next_row_tools_import_and_parse():
    """bench.py imports tools/bench_ndsp.py and tools/bench_dvbs2.py for its next_rows object: they must import without a GPU and accept the argument
    lists bench.py hands them (a failure inside them is recorded in the JSON line, but a syntax error would be silly to find on the GPU box)."""
    import importlib
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        nd = importlib.import_module("bench_ndsp")
        s2 = importlib.import_module("bench_dvbs2")
        a = nd.parse(["--steps", "4", "--warmup", "1", "--cpu-samples", "12000000"])
        b = s2.parse(["--rate", "2/3", "--sigma", "13"])
        assert a.samples == 1 << 30 and a.constellation == "qpsk" and b.front == 1 and b.sync_frames == 128 and b.esn0 == 8.0
        assert callable(nd.run) and callable(s2.run)
    finally:
        sys.path.remove(tools)
