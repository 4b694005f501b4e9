#!/bin/bash
# GPU box: the library with AddressSanitizer on the HOST code only (GPU ASan is not available on this pool), the runtime
# preloaded into Python, and the GPU suites that do not need torch (which does not initialise under the preload).
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
    -o py-numpy-renderer_amd/libmi355rast.so py-numpy-renderer_amd/csrc/mi355rast.hip || exit 1
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:log_path=$GRAFT_REPO_ROOT/gpurun_out/asan
rm -f gpurun_out/asan.*
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_overlay.py tests/test_host_api.py -q -m "gpu or not gpu" \
    -k "not torch and not BandRenderer and not frames_in_flight and not overflow and not split_frame and not band_renderer and not pipelined_stream" 2>&1 | tail -5
unset LD_PRELOAD
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
echo "ASan reports:"; ls gpurun_out/asan.* 2>/dev/null | wc -l; head -40 gpurun_out/asan.* 2>/dev/null
