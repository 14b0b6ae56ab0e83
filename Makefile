# Builds the product libraries in-tree (they travel to the GPU box with the gpurun snapshot):
#   diamond_amd/libdiamond_hip.so   HIP kernels + C ABI (include/diamond_hip.h), gfx950 only
#   diamond_amd/libdmnd_synth.so    deterministic synthetic workload generator (plain C)
# and the test-only helpers: oracle/_ref/* (CPU restatement + reference build), tests/emu/libswipe_emu.so
HIPCC   ?= /opt/rocm/bin/hipcc
CSRC    := diamond_amd/csrc
HIPSRC  := $(CSRC)/api.hip $(CSRC)/swipe_kernels.hip $(CSRC)/swipe16_kernels.hip $(CSRC)/seed_api.hip $(CSRC)/seed_kernels.hip $(CSRC)/extend_host.hip $(CSRC)/gapped_api.hip $(CSRC)/gapped_kernels.hip $(CSRC)/mask_api.hip $(CSRC)/mask_kernels.hip $(CSRC)/bias_kernels.hip $(CSRC)/format_api.hip $(CSRC)/frameshift_api.hip $(CSRC)/frameshift_kernels.hip $(CSRC)/frameshift_host.hip $(CSRC)/join_device.hip $(CSRC)/rank_join.hip $(CSRC)/plan_kernels.hip $(CSRC)/extend_kernels.hip $(CSRC)/extend_device.hip $(CSRC)/join_blocks.hip $(CSRC)/format_tab.hip $(CSRC)/translate.hip
HIPHDR  := $(wildcard $(CSRC)/*.h) include/diamond_hip.h
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function

.PHONY: all product oracle emu clean
all: product oracle emu

product: diamond_amd/libdiamond_hip.so diamond_amd/libdmnd_synth.so diamond_amd/diamond-hip motifs

# the reference's motif table for soft masking: generated where /root/reference exists, not committed
.PHONY: motifs
motifs:
	@python3 tools/make_motif_table.py >/dev/null || true

# one object per translation unit (no device code crosses a unit), so that a change to one kernel file rebuilds that file only
HIPOBJ  := $(patsubst $(CSRC)/%.hip,build/%.o,$(HIPSRC))
build/%.o: $(CSRC)/%.hip $(HIPHDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

# the device planner truncates double expressions to int exactly as the host's chaining does: no contraction into fused multiply-adds
build/plan_kernels.o build/extend_kernels.o: HIPFLAGS += -ffp-contract=off

# host-only double-precision code whose results must be bit-identical to the reference's (composition-based matrix adjustment):
# plain g++, no contraction of a * b + c into fused multiply-adds
HOSTOBJ := build/cbs_adjust.o
build/cbs_adjust.o: $(CSRC)/cbs_adjust.cpp $(HIPHDR)
	@mkdir -p build
	g++ -O2 -std=c++17 -ffp-contract=off -fPIC -Wall -c -o $@ $<

diamond_amd/libdiamond_hip.so: $(HIPOBJ) $(HOSTOBJ)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIPOBJ) $(HOSTOBJ) -ldl

# the CLI (makedb / blastp) over the C ABI; finds the library next to itself
diamond_amd/diamond-hip: $(CSRC)/cli.cpp include/diamond_hip.h diamond_amd/libdiamond_hip.so
	g++ -O2 -std=c++17 -Wall -o $@ $(CSRC)/cli.cpp -Ldiamond_amd -ldiamond_hip -lz -Wl,-rpath,'$$ORIGIN'

diamond_amd/libdmnd_synth.so: $(CSRC)/synth.c
	gcc -O2 -fPIC -shared -std=c11 -Wall -o $@ $< -lm

oracle:
	$(MAKE) -C oracle all

emu: tests/emu/libswipe_emu.so
tests/emu/libswipe_emu.so: $(wildcard tests/emu/*.cpp) $(wildcard $(CSRC)/*_core.h)
	g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -w -o $@ $(wildcard tests/emu/*.cpp)

clean:
	rm -f diamond_amd/*.so diamond_amd/diamond-hip tests/emu/*.so
	$(MAKE) -C oracle clean
