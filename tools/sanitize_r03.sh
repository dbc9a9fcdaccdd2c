#!/bin/bash
# The host C++ of the library (raftq_pipe.cpp, raftq_node.cpp, the host halves of the .hip units) under UBSan, ThreadSanitizer
# and AddressSanitizer, driven by the GPU suites that exercise it: the pipe (incl. its background thread), the node (chaos
# seeds, WAL restart, threaded cluster), the batching turn, the pipelined Step.  Run ON THE GPU BOX (gpurun); the sanitized
# libraries are built in-tree by raftsql_amd/build.py (build_sanitized) and travel with the snapshot.  VERDICT r02 item 6.
#   usage: tools/sanitize_r03.sh [outdir] [kinds...]      kinds default: ubsan tsan asan
set -u
P=${1:-gpurun_out/r03}; shift || true
KINDS=${*:-ubsan tsan asan}
mkdir -p $P
export TMPDIR=/tmp
cat > /tmp/tsan.supp <<'SUPP'
# uninstrumented runtimes: their internal synchronisation is invisible to TSan
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
called_from_lib:libtorch_hip.so
called_from_lib:libtorch_cpu.so
called_from_lib:libc10.so
race:libamdhip64.so
race:libhsa-runtime64.so
SUPP
TESTS="tests/test_pipe_gpu.py tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_parity_gpu.py::test_cycle_is_all_or_nothing_across_both_kinds tests/test_parity_gpu.py::test_cycle_zero_copy_staging tests/test_step_gpu.py::test_step_pipelined_submit_collect tests/test_wire_gpu.py::test_step_from_frames_staged_in_place"
for kind in $KINDS; do
  RT=$(python -c "from raftsql_amd import build as b; print(b.sanitizer_runtime('$kind'))")
  LOG=$P/sanitize_$kind.log
  rm -f $P/${kind}_report*
  echo "== $kind: LD_PRELOAD=$RT RAFTQ_LIB=raftsql_amd/libraftq_$kind.so ==" > $LOG
  case $kind in
    ubsan) env LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$P/ubsan_report RAFTQ_LIB=$PWD/raftsql_amd/libraftq_ubsan.so \
             timeout 1500 python -m pytest $TESTS -m gpu -q -p no:cacheprovider >> $LOG 2>&1 ;;
    tsan)  env LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=0:suppressions=/tmp/tsan.supp:log_path=$P/tsan_report:report_signal_unsafe=0 RAFTQ_LIB=$PWD/raftsql_amd/libraftq_tsan.so \
             timeout 1500 python -m pytest tests/test_pipe_gpu.py tests/test_node_gpu.py -m gpu -q -p no:cacheprovider -k "not host-memory" >> $LOG 2>&1 ;;
    asan)  # ROCm's ASan runtime intercepts the HSA allocator as well: the runtime's own pools must be allowed to fail over
           env LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:allocator_may_return_null=1:log_path=$P/asan_report HSA_XNACK=1 RAFTQ_LIB=$PWD/raftsql_amd/libraftq_asan.so \
             timeout 1500 python -m pytest $TESTS -m gpu -q -p no:cacheprovider -x >> $LOG 2>&1 ;;
  esac
  echo "rc=$?" >> $LOG
  ls $P/${kind}_report* >> $LOG 2>&1 || echo "no $kind report files: clean" >> $LOG
done
for f in $P/asan_report* $P/ubsan_report* $P/tsan_report*; do [ -f "$f" ] && { echo "---- $f"; head -60 "$f"; }; done > $P/sanitize_reports_head.txt 2>/dev/null
for kind in $KINDS; do echo "== $kind"; tail -n 4 $P/sanitize_$kind.log; done
