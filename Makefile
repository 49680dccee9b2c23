# Build of the MI355X phonon-transport engine and its host-side pieces.
#
#   make            host model builder + HIP engine + oracle (default)
#   make host       radiative3d_amd/lib/libr3d_host.so   (g++)
#   make engine     radiative3d_amd/lib/libr3d_hip.so    (hipcc, gfx950)
#   make oracle     oracle/libr3d_oracle.so              (g++; test infrastructure)
#   make cli        ./main                               (g++; the reference's command-line surface)
#
# `make -q` succeeding is what scripts/do-fundamentals.sh (reference :145-152)
# checks before a run.

CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
CXXFLAGS ?= -std=c++17 -O2 -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas
# -disable-machine-licm: the kernel is one long loop; hoisting every polynomial constant out of it
# costs ~60 registers (measured with machine-LICM on: 256 registers and 100+ spilled, DESIGN.md
# section 6); without the hoist the three kernels hold 162-167 and fit three waves per SIMD.
# -amdgpu-atomic-optimizer-strategy=None: the kernels' atomics on a uniform address are issued by ONE
# lane already (a batch's tallies, the id counter); the optimiser still wraps each in its generic
# reduction (lane election, a scalar loop over the active lanes): 1-2.5 % of the kernel time.
HIPFLAGS ?= -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical \
            -mllvm -disable-machine-licm -mllvm -amdgpu-atomic-optimizer-strategy=None
# The traversal kernels are one translation unit per cell kind (csrc/r3d_kernels_kind.hip), each with
# the instruction scheduler that suits it: under the compiler's max-ILP strategy the spherical-shell
# kernel runs 3.4 % faster (no vector register spilled, against 4) and the layered one 0.4 %, the
# tetra kernel 1.4 % slower (10 spilled against 4).
HIPFLAGS_CYL ?= -mllvm -amdgpu-sched-strategy=max-ilp
HIPFLAGS_TET ?=
HIPFLAGS_SPH ?= -mllvm -amdgpu-sched-strategy=max-ilp

LIBDIR   := radiative3d_amd/lib
HOSTDIR  := radiative3d_amd/host
CSRC     := radiative3d_amd/csrc

HOST_SRC := $(HOSTDIR)/ecs.cpp $(HOSTDIR)/grid.cpp $(HOSTDIR)/model.cpp \
            $(HOSTDIR)/models_builtin.cpp $(HOSTDIR)/cmdline.cpp $(HOSTDIR)/dataout.cpp \
            $(HOSTDIR)/capi.cpp
HOST_HDR := $(wildcard $(HOSTDIR)/*.hpp) include/r3d.h include/r3d_host.h
ENGINE_SRC := $(CSRC)/r3d_engine.hip $(CSRC)/r3d_tables_build.hip $(CSRC)/r3d_kernels_kind.hip $(CSRC)/r3d_volume.hip
ENGINE_HDR := $(wildcard $(CSRC)/*.h) include/r3d.h
OBJDIR   := build/obj

# (the default goal is the first target of the file: it has to come before the generated object rules)
.PHONY: default all host engine repro oracle cli clean
default: all
# engine_objects(tag, extra flags): the six objects of one engine build
define engine_objects
$(OBJDIR)/$(1)_engine.o: $(CSRC)/r3d_engine.hip $(ENGINE_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(2) -c -o $$@ $(CSRC)/r3d_engine.hip
$(OBJDIR)/$(1)_tables.o: $(CSRC)/r3d_tables_build.hip $(ENGINE_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(2) -c -o $$@ $(CSRC)/r3d_tables_build.hip
$(OBJDIR)/$(1)_volume.o: $(CSRC)/r3d_volume.hip $(ENGINE_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(2) -c -o $$@ $(CSRC)/r3d_volume.hip
$(OBJDIR)/$(1)_cyl.o: $(CSRC)/r3d_kernels_kind.hip $(ENGINE_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_CYL) $(2) -DR3D_KIND=0 -c -o $$@ $(CSRC)/r3d_kernels_kind.hip
$(OBJDIR)/$(1)_tet.o: $(CSRC)/r3d_kernels_kind.hip $(ENGINE_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_TET) $(2) -DR3D_KIND=1 -c -o $$@ $(CSRC)/r3d_kernels_kind.hip
$(OBJDIR)/$(1)_sph.o: $(CSRC)/r3d_kernels_kind.hip $(ENGINE_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_SPH) $(2) -DR3D_KIND=2 -c -o $$@ $(CSRC)/r3d_kernels_kind.hip
endef
engine_objs = $(OBJDIR)/$(1)_engine.o $(OBJDIR)/$(1)_tables.o $(OBJDIR)/$(1)_volume.o $(OBJDIR)/$(1)_cyl.o $(OBJDIR)/$(1)_tet.o $(OBJDIR)/$(1)_sph.o
$(eval $(call engine_objects,main,))
$(eval $(call engine_objects,repro,-DR3D_REPRODUCIBLE -ffp-contract=off))

all: host engine repro oracle cli

host: $(LIBDIR)/libr3d_host.so
engine: $(LIBDIR)/libr3d_hip.so
# the same engine with every wave-voted series choice taken out (r3d_math.h all_lanes) and no
# contraction of a * b + c into fused multiply-adds beyond the ones the sources spell out (the
# compiler contracts differently in each kernel it inlines the physics into: the diagnostic and the
# production kernels then differ in the last bits, ~1e-10 after an ill-conditioned travel time): a
# history's result is then bit-defined by (model, seed, id) in every kernel; loaded under
# R3D_REPRODUCIBLE=1.  Costs 5-10 %.
repro: $(LIBDIR)/libr3d_hip_repro.so
oracle: oracle/libr3d_oracle.so oracle/libr3d_tables_oracle.so
cli: main

# The command-line program the reference's do-*.sh scripts call as ./main
main: $(HOSTDIR)/main.cpp $(LIBDIR)/libr3d_host.so $(LIBDIR)/libr3d_hip.so $(ENGINE_HDR)
	$(CXX) $(CXXFLAGS) -pthread -o $@ $(HOSTDIR)/main.cpp -L$(LIBDIR) -lr3d_host -lr3d_hip \
	    -Wl,-rpath,'$$ORIGIN/$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

$(LIBDIR)/libr3d_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -pthread -o $@ $(HOST_SRC)

# (librccl is bound at first use, csrc/r3d_rccl.h: the copy already in the process if there is one)
RCCL_LIBS ?= -ldl
$(LIBDIR)/libr3d_hip.so: $(call engine_objs,main)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -pthread -o $@ $(call engine_objs,main) $(RCCL_LIBS)

$(LIBDIR)/libr3d_hip_repro.so: $(call engine_objs,repro)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -pthread -o $@ $(call engine_objs,repro) $(RCCL_LIBS)

oracle/libr3d_oracle.so: oracle/r3d_oracle.cpp oracle/philox.h include/r3d.h
	$(CXX) $(CXXFLAGS) -shared -o $@ oracle/r3d_oracle.cpp

# the table builders restated (take-off set, scatterer tables, source patterns, seismometer axes):
# plain g++, no fast-math, no FMA contraction -- the reference's operation order is the point
oracle/libr3d_tables_oracle.so: oracle/r3d_tables_oracle.cpp
	$(CXX) -std=c++17 -O2 -ffp-contract=off -fPIC -Wall -shared -o $@ oracle/r3d_tables_oracle.cpp

clean:
	rm -f $(LIBDIR)/*.so oracle/*.so main
	rm -rf $(OBJDIR)

# Developer variants of the engine (timing-only / diagnostic builds, never shipped as libr3d_hip.so;
# the only builds that say -DR3D_DEV_BUILD, without which the R3D_ABLATE_* / R3D_PHASE_TIMING switches
# are a compile error -- csrc/r3d_tables.h):
#   make variant NAME=PHASE DEFS="-DR3D_PHASE_TIMING"   ->  radiative3d_amd/lib/variant_PHASE.so
# run through the tools with R3D_HIP_LIB=radiative3d_amd/lib/variant_PHASE.so (tools/time_chain.py,
# tools/pool_stats.py pass it to Engine(lib=...); the library itself reads no environment).
# Every object is compiled afresh and each compile's status is checked: a failed one cannot leave
# an older object to be linked in its place.
VOBJ = $(OBJDIR)/v$(NAME)
VDEFS = -DR3D_DEV_BUILD $(DEFS)
variant:
	@mkdir -p $(OBJDIR)
	rm -f $(VOBJ)_engine.o $(VOBJ)_tables.o $(VOBJ)_volume.o $(VOBJ)_cyl.o $(VOBJ)_tet.o $(VOBJ)_sph.o $(LIBDIR)/variant_$(NAME).so
	$(HIPCC) $(HIPFLAGS) $(VDEFS) -c -o $(VOBJ)_engine.o $(CSRC)/r3d_engine.hip & p1=$$!; \
	$(HIPCC) $(HIPFLAGS) $(VDEFS) -c -o $(VOBJ)_tables.o $(CSRC)/r3d_tables_build.hip & p2=$$!; \
	$(HIPCC) $(HIPFLAGS) $(VDEFS) -c -o $(VOBJ)_volume.o $(CSRC)/r3d_volume.hip & p6=$$!; \
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_CYL) $(VDEFS) $(DEFS_CYL) -DR3D_KIND=0 -c -o $(VOBJ)_cyl.o $(CSRC)/r3d_kernels_kind.hip & p3=$$!; \
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_TET) $(VDEFS) $(DEFS_TET) -DR3D_KIND=1 -c -o $(VOBJ)_tet.o $(CSRC)/r3d_kernels_kind.hip & p4=$$!; \
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_SPH) $(VDEFS) $(DEFS_SPH) -DR3D_KIND=2 -c -o $(VOBJ)_sph.o $(CSRC)/r3d_kernels_kind.hip & p5=$$!; \
	rc=0; for p in $$p1 $$p2 $$p3 $$p4 $$p5 $$p6; do wait $$p || rc=1; done; exit $$rc
	$(HIPCC) --offload-arch=gfx950 -shared -pthread -o $(LIBDIR)/variant_$(NAME).so $(VOBJ)_engine.o \
	    $(VOBJ)_tables.o $(VOBJ)_volume.o $(VOBJ)_cyl.o $(VOBJ)_tet.o $(VOBJ)_sph.o $(RCCL_LIBS)
