# Convenience targets (the driver contract is __graft_entry__.py + bench.py; these only wrap them)
PY ?= python

build:            ## hipcc --offload-arch=gfx950: deepsignal_plant_amd/libdsp_amd.so (+ trace build, oracle)
	$(PY) -c 'import __graft_entry__ as g; g.build()'

test:             ## CPU suite (oracle vs fixtures, host code, fast5 reader, sharding)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## GPU suite (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

smoke:            ## one small forward + extraction + call_freq on cuda:0 against the oracle
	$(PY) -c 'import __graft_entry__ as g; g.smoke()'

bench:            ## config 1, one JSON line
	$(PY) bench.py

fixtures:         ## regenerate the reference fixtures (this container only: reads /root/reference)
	$(PY) tests/golden/make_golden.py
	$(PY) tests/golden/make_golden_text.py
	$(PY) tests/golden/make_golden_extract.py
	/opt/conda/bin/python3.9 tests/golden/make_golden_fast5.py

.PHONY: build test test-gpu smoke bench fixtures
