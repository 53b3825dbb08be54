# Builds libdgv2.so (HIP kernels for gfx950 behind the C ABI of include/dgv2.h).  hipcc cross-compiles without a GPU.
# `make ABLATE=1` compiles the benchmark-only ablation switches (DGV2_*_ABLATE: skip stores / MFMA loops; WRONG results)
# into the kernels; the default build has none of them.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := dusty-gan-v2_amd
SRC := $(wildcard $(PKG)/csrc/*.hip)
OBJ := $(patsubst $(PKG)/csrc/%.hip,build/%.o,$(SRC))
LIB := $(PKG)/lib/libdgv2.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude $(if $(ABLATE),-DDGV2_ABLATE) $(if $(FIR_RS),-DDGV2_FIR_RS=$(FIR_RS))

all: $(LIB)

build/%.o: $(PKG)/csrc/%.hip $(wildcard $(PKG)/csrc/*.h) include/dgv2.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
