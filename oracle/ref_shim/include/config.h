/* oracle/ref_shim: stands in for the autoconf-generated config.h of the reference
 * (configure.ac:1 -> version 0.6.5).  TEST INFRASTRUCTURE ONLY. */
#pragma once
#define VERSION "0.6.5"
#define PACKAGE_VERSION "0.6.5"
#define HAVE_FFMPEG 0
