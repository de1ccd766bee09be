# Top-level build: host model layer, HIP engine, command line driver; `make oracle` builds the test oracle.
# No cmake: plain g++ / hipcc (gfx950 only).
CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
# host code generation matches the reference's Release build (-O3, no -march => no FMA contraction) so that
# setup-time tables are bit-identical to SKIRT's
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter
HIPFLAGS_CODE := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics
# (the flags that decide the generated code are recorded in the binary: pmc_build_info)
HIPFLAGS := $(HIPFLAGS_CODE) -Wall -Wno-unused-parameter -Wno-unused-value -DPMC_BUILD_FLAGS='"$(HIPFLAGS_CODE)"'

HOST_SRC := $(wildcard skirt9_amd/host/*.cpp)
HOST_HDR := $(wildcard skirt9_amd/host/*.hpp) $(wildcard include/*.h)
HOST_LIB_SRC := $(filter-out skirt9_amd/host/main.cpp,$(HOST_SRC))

all: skirt9_amd/lib/libskirthost.so skirt9_amd/lib/libpmc.so skirt9_amd/lib/skirt_mi355x

skirt9_amd/lib/libskirthost.so: $(HOST_LIB_SRC) $(HOST_HDR)
	@mkdir -p skirt9_amd/lib
	$(CXX) $(CXXFLAGS) -pthread -shared $(HOST_LIB_SRC) -o $@

skirt9_amd/lib/libpmc.so: $(wildcard skirt9_amd/csrc/*.hip) $(wildcard skirt9_amd/csrc/*.h) $(wildcard skirt9_amd/csrc/*.inc) $(wildcard include/*.h)
	@mkdir -p skirt9_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared $(wildcard skirt9_amd/csrc/*.hip) -lrccl -o $@

# tuning aid: the same engine with in-kernel cycle stamps (PMC_LIBRARY=.../libpmc_prof.so PMC_PROFILE_DUMP=1)
profile-lib: skirt9_amd/lib/libpmc_prof.so
skirt9_amd/lib/libpmc_prof.so: $(wildcard skirt9_amd/csrc/*.hip) $(wildcard skirt9_amd/csrc/*.h) $(wildcard skirt9_amd/csrc/*.inc) $(wildcard include/*.h)
	@mkdir -p skirt9_amd/lib
	$(HIPCC) $(HIPFLAGS) -DPMC_PROFILE -DPMC_PROFILE_STAMPS -shared $(wildcard skirt9_amd/csrc/*.hip) -lrccl -o $@

# event census of the walk kernels without the time stamps (PMC_LIBRARY=.../libpmc_census.so PMC_PROFILE_DUMP=1)
census-lib: skirt9_amd/lib/libpmc_census.so
skirt9_amd/lib/libpmc_census.so: $(wildcard skirt9_amd/csrc/*.hip) $(wildcard skirt9_amd/csrc/*.h) $(wildcard skirt9_amd/csrc/*.inc) $(wildcard include/*.h)
	@mkdir -p skirt9_amd/lib
	$(HIPCC) $(HIPFLAGS) -DPMC_PROFILE -shared $(wildcard skirt9_amd/csrc/*.hip) -lrccl -o $@

skirt9_amd/lib/skirt_mi355x: skirt9_amd/host/main.cpp skirt9_amd/lib/libskirthost.so skirt9_amd/lib/libpmc.so
	$(CXX) $(CXXFLAGS) skirt9_amd/host/main.cpp -Lskirt9_amd/lib -lskirthost -lpmc -pthread -Wl,-rpath,'$$ORIGIN' -o $@

# ---- test oracle (never part of `all`)
oracle: oracle/_build/liboracle.so
oracle/_build/liboracle.so: oracle/life_cycle.cpp $(wildcard include/*.h)
	@mkdir -p oracle/_build
	$(CXX) $(CXXFLAGS) -shared oracle/life_cycle.cpp -o $@

host: skirt9_amd/lib/libskirthost.so
clean:
	rm -rf skirt9_amd/lib oracle/_build
.PHONY: all oracle host clean

# tuning aid: a variant of the engine with extra definitions (tools/sweep.py picks it by file name), e.g.
#   make variant NAME=pert_valu_48 DEFS=-DPMC_PERTURB_VALU=48     -> skirt9_amd/lib/libpmc_pert_valu_48.so
variant:
	@mkdir -p skirt9_amd/lib
	$(HIPCC) $(HIPFLAGS) $(DEFS) -shared $(wildcard skirt9_amd/csrc/*.hip) -lrccl -o skirt9_amd/lib/libpmc_$(NAME).so
.PHONY: variant
