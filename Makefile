# Builds libdgv2.so (HIP kernels for gfx950 behind the C ABI of include/dgv2.h).  hipcc cross-compiles without a GPU.
# `make ABLATE=1` compiles the benchmark-only ablation switches (DGV2_*_ABLATE: skip stores / MFMA loops; WRONG results)
# into the kernels; the default build has none of them.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := dusty-gan-v2_amd
SRC := $(wildcard $(PKG)/csrc/*.hip)
# `make BUILD=build_x LIB=build_x/libdgv2.so EXTRA=-DDGV2_...` builds an experiment variant beside the shipped library
# (DGV2_LIB_PATH points the binding at it for same-box A/B runs).
BUILD ?= build
OBJ := $(patsubst $(PKG)/csrc/%.hip,$(BUILD)/%.o,$(SRC))
LIB ?= $(PKG)/lib/libdgv2.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude $(if $(ABLATE),-DDGV2_ABLATE) $(if $(FIR_RS),-DDGV2_FIR_RS=$(FIR_RS)) $(EXTRA)

all: $(LIB)

$(BUILD)/%.o: $(PKG)/csrc/%.hip $(wildcard $(PKG)/csrc/*.h) include/dgv2.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf $(BUILD) $(LIB)

.PHONY: all clean
