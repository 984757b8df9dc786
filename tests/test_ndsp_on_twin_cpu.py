"""The ndsp GPU parity tests (tests/test_ndsp_gpu.py), the same functions, collected against the HOST TWIN of the engine (tests/emu) in
the CPU suite: host logic and arithmetic of the ndsp chain are proven here; the GPU build stays with -m gpu."""
from tests import test_ndsp_gpu as N
from tests.test_demod_gpu_on_twin_cpu import capi, torch_cuda  # noqa: F401  (fixtures: the twin's binding, numpy stand-in for torch)

nref = N.nref

test_ndsp_blocks_bit_exact = N.test_ndsp_blocks_bit_exact
test_ndsp_psk_demod_exact_bit_identical = N.test_ndsp_psk_demod_exact_bit_identical
test_ndsp_psk_demod_advanced_keys = N.test_ndsp_psk_demod_advanced_keys
test_ndsp_psk_demod_chunk_parallel = N.test_ndsp_psk_demod_chunk_parallel
test_ndsp_psk_demod_golden = N.test_ndsp_psk_demod_golden
test_ndsp_host_mirror = N.test_ndsp_host_mirror
test_ndsp_single_block_handles = N.test_ndsp_single_block_handles
test_ndsp_agc_scan_start_gains = N.test_ndsp_agc_scan_start_gains
test_ndsp_costas_fast_chunk_parallel = N.test_ndsp_costas_fast_chunk_parallel
test_ndsp_mm_fast_chunk_parallel = N.test_ndsp_mm_fast_chunk_parallel
test_ndsp_mm_fast_short_warmup = N.test_ndsp_mm_fast_short_warmup
