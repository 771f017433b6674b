# Convenience targets; the driver uses __graft_entry__.build().
PY ?= python

all:
	$(PY) -c "import __graft_entry__ as g; g.build()"

# Host-side ASan + UBSan builds (CPU box only): the C-ABI library with the host code of every translation unit
# instrumented (deblurgs_amd/libdgs_hip_san.so; device code untouched) and the CPU oracle (oracle/libdgs_oracle_san.so).
# tests/test_sanitize.py builds both and runs the ABI / oracle-golden tests against them under the sanitizer runtimes.
sanitize:
	$(PY) -m deblurgs_amd.build --sanitize
	$(MAKE) -C oracle sanitize

test-sanitize: sanitize
	$(PY) -m pytest tests/test_sanitize.py -q

.PHONY: all sanitize test-sanitize
