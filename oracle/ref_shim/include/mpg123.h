/* oracle/ref_shim: opaque handle only; mp3inputstream.cc is NOT compiled into the
 * oracle (MP3 is out of scope). */
#pragma once
typedef struct mpg123_handle_struct mpg123_handle;
