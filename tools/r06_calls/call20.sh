#!/bin/bash
# round 6 call 20: render() of one camera under no_grad on the raw-parameter path: the new test and the tests around it
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -q -k "render_of_one_camera or forward_only or debug_mode_and_render or l0_C or fused_step_equals or toy_deblurring" 2>&1 | tail -12 | cut -c1-220
