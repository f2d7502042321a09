# Recipe for the pieces of the REAL reference that compile from their own few source files
# with plain gcc (no configure, no generated headers).  Outputs go to oracle/_ref/ only
# (git-ignored).  Runs only where /root/reference exists (the build container).
#
#   videogen  -- tests/videogen.c (+ tests/utils.c, #included by it): the reference's own
#                synthetic test-video generator that FATE uses for the swscale goldens
#                (tests/Makefile:29-38, tests/fate/libswscale.mak).
#
# libswscale itself is NOT buildable this way: every source includes the configure-generated
# config.h / config_components.h / libavutil/avconfig.h / ffversion.h.
REF ?= /root/reference
OUT := _ref

all: $(OUT)/videogen

$(OUT)/videogen: $(REF)/tests/videogen.c $(REF)/tests/utils.c
	mkdir -p $(OUT)
	gcc -O2 -I$(REF) -o $@ $(REF)/tests/videogen.c
