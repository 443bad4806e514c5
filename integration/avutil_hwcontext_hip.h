/* integration/avutil_hwcontext_hip.h — libavutil/hwcontext_hip.h of the FFmpeg-side patch: the public part of the `hip` device type
 * (the counterpart of libavutil/hwcontext_cuda.h:40-65).  AVHWDeviceContext.hwctx of a hip device is an AVHIPDeviceContext; frames of
 * a hip AVHWFramesContext carry device pointers in data[] and byte strides in linesize[], format FFHIP_HW_PIX_FMT. */
#ifndef FFHIP_INTEGRATION_HWCONTEXT_HIP_H
#define FFHIP_INTEGRATION_HWCONTEXT_HIP_H

#include "libavutil/hwcontext.h"
#include "libavutil/pixfmt.h"

/* the real patch adds AV_HWDEVICE_TYPE_HIP / AV_PIX_FMT_HIP; this build borrows the CUDA slot of the unmodified hwcontext.c
 * (see avutil_hwcontext_hip.c and avutil_hwcontext_table_hip.c) */
#define FFHIP_HWDEVICE_TYPE AV_HWDEVICE_TYPE_CUDA
#define FFHIP_HW_PIX_FMT    AV_PIX_FMT_CUDA

typedef struct AVHIPDeviceContext {
    int   device;        /* HIP device ordinal: contexts of libffhip created while it is current are bound to it */
    void *stream;        /* hipStream_t of the device: transfers run on it; hand it to the ffhip_*_dev calls that consume the frames */
    int   owns_stream;
    int   async_upload;  /* set by the user: uploads return without waiting (the consumer is ordered behind them on `stream`) */
} AVHIPDeviceContext;

#endif
