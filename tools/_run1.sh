export TMPDIR=/tmp PYTHONPATH=$PWD
python tools/sort_rays_ab.py 2>/dev/null | tail -1
python tools/aten_time.py --top 40 2>/dev/null | grep "aten::copy_ " | head -8 | cut -c1-160
