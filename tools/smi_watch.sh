#!/bin/bash
# Diagnostics: sample the GPU's clock and power (rocm-smi) while a command runs.  usage: smi_watch.sh OUT.log -- command ...
OUT=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' ' ; echo; sleep 0.2; done ) > "$OUT" &
W=$!
"$@"
RC=$?
kill $W 2>/dev/null
exit $RC
