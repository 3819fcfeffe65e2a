cd /root/repo
python tools/dbg_ml.py 9103 sub 2>&1 | cut -c1-700 | tail -40
