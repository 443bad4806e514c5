/* error.c — thread-local last-error text for the C-ABI (the reference reports through av_log();
 * a C-ABI boundary cannot, so the text is kept for ffhip_last_error()). */
#include <stdarg.h>
#include <stdio.h>
#include "ffhip_internal.h"

static _Thread_local char last_error[512];

void ffhip_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error, sizeof(last_error), fmt, ap);
    va_end(ap);
}

const char *ffhip_last_error(void) { return last_error; }
const char *ffhip_version(void) { return "ffhip 0.1 (gfx950)"; }
