#!/usr/bin/env python3
"""Drop-in command for `python BalLeRMix+_v1.py ...` (same flags) running the scan on MI355X."""
from ballermixplus_amd.cli import main

if __name__ == '__main__':
    main()
