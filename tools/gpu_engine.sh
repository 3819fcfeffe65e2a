#!/bin/bash
# where a 2 MB chunk's time goes inside the reference's flb_processor_run with the GPU plugins as its units: each unit alone, and the built-in grep alone
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import synth
data, off, ep = synth.apache_records(7000)
open("/tmp/in.mp", "wb").write(bytes(data[: int(off[7000])]))
PY
E=oracle/_ref/engine
P='apache2|^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$|%d/%b/%Y:%H:%M:%S %z|time|0'
G="-e $E/plugins/flb-filter_parser_gpu.so -e $E/plugins/flb-filter_grep_gpu.so"
run() { echo "== $1"; shift; timeout 300 $E/engine_host processor "$@" 2>&1 | grep '^{' | cut -c1-400; }
run "parser_gpu alone" $G --parser "$P" --repeat 200 /tmp/in.mp /tmp/o1 --unit parser_gpu key_name=log parser=apache2
run "grep_gpu alone (keeps nothing of the raw lines: regex on log)" $G --repeat 200 /tmp/in.mp /tmp/o2 --unit grep_gpu 'regex=log HTTP/1.1" 5'
run "grep_gpu alone, NOTOUCH (keeps everything)" $G --repeat 200 /tmp/in.mp /tmp/o3 --unit grep_gpu 'regex=log .'
run "built-in grep alone, NOTOUCH" --repeat 200 /tmp/in.mp /tmp/o4 --unit grep 'regex=log .'
run "built-in grep alone, drops" --repeat 200 /tmp/in.mp /tmp/o5 --unit grep 'regex=log HTTP/1.1" 5'
run "both gpu units" $G --parser "$P" --repeat 200 /tmp/in.mp /tmp/o6 --unit parser_gpu key_name=log parser=apache2 --unit grep_gpu 'regex=code ^5\d\d$'
