# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
PY ?= python

build:            ## librc_hip.so + rc_render (hipcc, gfx950) and the checker (oracle/)
	$(PY) -m cudaraytracing_amd.build
	$(MAKE) -C oracle all

test-cpu: build   ## everything that needs no GPU
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## on an MI355X, from a tree that already holds the built libraries
	$(PY) -u -m pytest tests -q -m gpu --timeout 300 --timeout-method thread

bench:
	$(PY) bench.py

clean:
	rm -f cudaraytracing_amd/librc_hip.so cudaraytracing_amd/rc_render
	$(MAKE) -C oracle clean

.PHONY: build test-cpu test-gpu bench clean
