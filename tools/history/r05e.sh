for v in "HTA_DBG_PRINT=1" "HTA_DBG_LDS_EXTRA=16384"; do echo "== $v"; env R05D_ONLY100=1 $v python tools/history/r05d.py 2>&1 | grep "float64 100\|metric_eval:" | head -12 | cut -c1-200; done
