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

# Every object is assembled from the ISA the audit below has read (-save-temps=obj keeps that .s next to the object):
# hipcc's hazard recogniser does not look inside `asm`, so a compiler-generated VALU write into a register an asm-issued
# MFMA is still reading -- or a read of its result in flight -- goes unpadded.
# scripts/audit_asm_mfma.py scans the ISA for exactly that and FAILS THE BUILD on a finding, whatever the flags
# (EXTRA=, ABLATE=1, another ARCH, a compiler upgrade); sources without asm MFMAs pass trivially.
PYTHON ?= python3
$(BUILD)/%.o: $(PKG)/csrc/%.hip $(wildcard $(PKG)/csrc/*.h) include/dgv2.h scripts/audit_asm_mfma.py
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -save-temps=obj -c $< -o $@
	@$(PYTHON) scripts/audit_asm_mfma.py $(BUILD)/$*-hip-amdgcn-amd-amdhsa-$(ARCH).s > $(BUILD)/$*.audit \
	  || { cat $(BUILD)/$*.audit; echo "asm-MFMA hazard audit FAILED for $<"; rm -f $@; exit 1; }
	@rm -f $(BUILD)/$*-hip-*.bc $(BUILD)/$*-hip-*.hipi $(BUILD)/$*-hip-*.out $(BUILD)/$*-hip-*.out.resolution.txt \
	  $(BUILD)/$*-host-*.bc $(BUILD)/$*-host-*.hipi $(BUILD)/$*-host-*.s $(BUILD)/$*.hip-hip-*.hipfb $(BUILD)/$*-hip-*.o

$(LIB): $(OBJ)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf $(BUILD) $(LIB)

.PHONY: all clean
