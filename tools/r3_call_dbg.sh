cd /root/repo
python tools/dbg_ml.py 5201 2>&1 | head -40
