bash /root/repo/tools/r3_call_ml.sh
bash /root/repo/tools/r3_call_mlperf.sh
