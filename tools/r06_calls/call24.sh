#!/bin/bash
python tests/fuzz_case_report.py 604 172 128 5 9085 0.7495761496122001 1 use_sigmoid 2>&1 | tail -22
python tests/fuzz_case_report.py 2550 107 116 1 9029 0.8903742826575616 1 use_sigmoid 2>&1 | tail -16
