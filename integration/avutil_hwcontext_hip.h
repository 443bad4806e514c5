/* integration/avutil_hwcontext_hip.h — libavutil/hwcontext_hip.h of the FFmpeg-side patch: the public part of the `hip` device type
 * (the counterpart of libavutil/hwcontext_cuda.h:40-65).  AVHWDeviceContext.hwctx of a hip device is an AVHIPDeviceContext; frames of
 * a hip AVHWFramesContext carry device pointers in data[] and byte strides in linesize[], format FFHIP_HW_PIX_FMT. */
#ifndef FFHIP_INTEGRATION_HWCONTEXT_HIP_H
#define FFHIP_INTEGRATION_HWCONTEXT_HIP_H

#include "libavutil/hwcontext.h"
#include "libavutil/pixfmt.h"

/* The patch appends one enumerator to each public enum: AV_HWDEVICE_TYPE_HIP after AV_HWDEVICE_TYPE_OHCODEC (libavutil/hwcontext.h:43)
 * and AV_PIX_FMT_HIP before AV_PIX_FMT_NB (libavutil/pixfmt.h:508).  The reference headers are compiled as they lie, so here the
 * two are the same VALUES spelled as constants: the first free device type and the first free pixel format.  The files that size or
 * initialise a table by these enums get the extra row from a wrapper that compiles the reference file where it lies:
 * hw_type_names[] / hw_table[] in avutil_hwcontext_table_hip.c, av_pix_fmt_descriptors[] in avutil_pixdesc_hip.c. */
enum { FFHIP_PIX_FMT_NB_REF = AV_PIX_FMT_NB, FFHIP_HWDEVICE_TYPE_NB_REF = AV_HWDEVICE_TYPE_OHCODEC + 1 };
#define AV_HWDEVICE_TYPE_HIP ((enum AVHWDeviceType)FFHIP_HWDEVICE_TYPE_NB_REF)
#define AV_PIX_FMT_HIP       ((enum AVPixelFormat)FFHIP_PIX_FMT_NB_REF)
#define FFHIP_HWDEVICE_TYPE  AV_HWDEVICE_TYPE_HIP   /* (the names the integration files were written with) */
#define FFHIP_HW_PIX_FMT     AV_PIX_FMT_HIP

typedef struct AVHIPDeviceContext {
    int   device;        /* HIP device ordinal: contexts of libffhip created while it is current are bound to it */
    void *stream;        /* hipStream_t of the device: transfers run on it; hand it to the ffhip_*_dev calls that consume the frames */
    int   owns_stream;
    int   async_upload;  /* set by the user: uploads return without waiting (the consumer is ordered behind them on `stream`) */
} AVHIPDeviceContext;

#endif
