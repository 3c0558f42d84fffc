# cmake -P script (NOT the reference's build system): instantiates libdivsufsort's own header templates
#   /root/reference/ext/libdivsufsort/include/{divsufsort.h.cmake,config.h.cmake}
# with configure_file(), the way ext/libdivsufsort/include/CMakeLists.txt does on x86-64 Linux / gcc
# (uint8_t, int32_t / int64_t, PRId32 / PRId64, inline; its CheckTypeSize / CheckFunctionKeywords probes
# all succeed on this platform, see include/CMakeLists.txt:58-160).  Output goes to oracle/_ref/include only.
#   cmake -DSRC=<libdivsufsort dir> -DOUT=<oracle/_ref/include> -P divsufsort_headers.cmake
if(NOT SRC OR NOT OUT)
  message(FATAL_ERROR "usage: cmake -DSRC=... -DOUT=... -P divsufsort_headers.cmake")
endif()
file(STRINGS "${SRC}/VERSION" DSS_VERSION LIMIT_COUNT 1)
set(PROJECT_VERSION_FULL "${DSS_VERSION}")
foreach(h INTTYPES STDDEF STDINT STDLIB STRING STRINGS MEMORY SYS_TYPES)
  set(HAVE_${h}_H 1)
endforeach()
set(INLINE "inline")
configure_file("${SRC}/include/config.h.cmake" "${OUT}/config.h")
set(INCFILE "#include <inttypes.h>")
set(DIVSUFSORT_IMPORT "")
set(DIVSUFSORT_EXPORT "")
set(SAUCHAR_TYPE "uint8_t")
set(SAINT32_TYPE "int32_t")
set(SAINT_PRId "PRId32")
set(W64BIT "")
set(SAINDEX_TYPE "int32_t")
set(SAINDEX_PRId "PRId32")
configure_file("${SRC}/include/divsufsort.h.cmake" "${OUT}/divsufsort.h" @ONLY)
set(W64BIT "64")
set(SAINDEX_TYPE "int64_t")
set(SAINDEX_PRId "PRId64")
configure_file("${SRC}/include/divsufsort.h.cmake" "${OUT}/divsufsort64.h" @ONLY)
