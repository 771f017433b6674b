#!/bin/bash
# round 6 call 28: the tests that use the (now chunked) _pair_can_contribute helper, plus the other tests of that file that the
# sweep refactor touched -- small scenes only
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tile_cull or bit_exact or fuzz_shapes" -p no:cacheprovider 2>&1 | tail -4
