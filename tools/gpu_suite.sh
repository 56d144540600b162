#!/bin/bash
# Full GPU validation on a box: the -m gpu suite and smoke().  The exit status is pytest's (a time-out or a failure is not hidden
# behind the tail of the log); the limit leaves room for a cold hipcc build of the libraries inside the first test.
set -o pipefail
mkdir -p gpurun_out/suite
(time timeout ${SUITE_TIMEOUT:-2400} python -m pytest tests -q -m gpu) 2>&1 | tail -${SUITE_TAIL:-12} | tee gpurun_out/suite/gpu_tests.log
rc=$?
echo "pytest exit status: $rc" | tee -a gpurun_out/suite/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/suite/smoke.log
exit $rc
