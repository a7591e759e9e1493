# convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly)
.PHONY: build test test-gpu bench golden clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu
bench:
	python bench.py
golden:            # only where /root/reference is mounted (build container): ALL three fixture sets (tests/test_cpu_golden_regenerates.py re-runs them into a scratch directory)
	python tests/golden/make_golden.py
	python tests/golden/make_golden_net.py
	python tests/golden/make_golden_game.py
clean:
	$(MAKE) -C chinesechesszero_amd/csrc clean
	$(MAKE) -C oracle clean
