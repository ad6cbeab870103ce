# Top-level Makefile -- the same build as `python -c "import __graft_entry__ as g; g.build()"`, for C / C++ users without Python:
#   make            libdil256.so (HIP kernels + C-ABI, gfx950) and libdil256_ref.so (the reference-identical C++ symbols)
#   make checkers   the oracle (test infrastructure) and, where /root/reference exists, the compiled reference under oracle/_ref/
#   make cpp-tests  the C++ mains under tests/cpp/ (linked against the two libraries; they need a GPU to run)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CSRC    := dilithium_amd/csrc
HIP_SRC := $(addprefix $(CSRC)/,kernels.hip pipelines.hip hash_kernels.hip coop_kernels.hip codec_kernels.hip wire_kernels.hip capi.hip scheme.hip multi_gpu.hip)
HIP_HDR := $(wildcard $(CSRC)/*.hpp) include/dil256.h include/dil256_ref.hpp

all: dilithium_amd/libdil256.so dilithium_amd/libdil256_ref.so

OBJ_DIR := dilithium_amd/build
HIP_OBJ := $(patsubst $(CSRC)/%.hip,$(OBJ_DIR)/%.o,$(HIP_SRC))

$(OBJ_DIR)/%.o: $(CSRC)/%.hip $(HIP_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -pthread -c $< -o $@

# one object per translation unit: `make -j` compiles them side by side (the whole library: ~15 s instead of ~45)
dilithium_amd/libdil256.so: $(HIP_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -pthread $(HIP_OBJ) -o $@

dilithium_amd/libdil256_ref.so: $(CSRC)/ref_api.cpp dilithium_amd/libdil256.so include/dil256_ref.hpp
	$(CXX) -O2 -std=c++17 -shared -fPIC -Wall $< -Ldilithium_amd -ldil256 -Wl,-rpath,'$$ORIGIN' -o $@

checkers:
	$(MAKE) -C oracle
ifeq ($(SAN),1)
	$(MAKE) -C oracle san
endif

# AddressSanitizer + UBSan over the CPU side: the oracle, the drop-in's host code (ref_api.cpp) and the HOST code of the runtime
# (capi.hip, scheme.hip arenas / options, multi_gpu.hip) -- two passes, one per sanitizer runtime (gcc's for the C / C++ files,
# clang's for the hipcc-compiled ones); logs under profiles/.   make sanitize
sanitize:
	bash scripts/san_check.sh

cpp-tests: all
	$(MAKE) -C tests/cpp

.PHONY: all checkers cpp-tests sanitize
