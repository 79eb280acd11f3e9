"""bench.py's parts: models (pure roofline arithmetic), workloads, cpu (the CPU baseline leg), measure, line (the compact record)."""
