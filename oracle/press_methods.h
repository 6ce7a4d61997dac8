/*
 * Method ids shared by the oracle (CPU restatement), the reference shim
 * (oracle/_ref) and the HIP library (include/press_hip.h uses the same values).
 * TEST INFRASTRUCTURE ONLY - nothing under oracle/ is linked into the product.
 *
 * Names are the reference's code names (press/press.h); thesis names in comments.
 */
#ifndef PRESS_METHODS_H
#define PRESS_METHODS_H

enum press_method {
	PM_SVB12            = 0,  /* svb16, no zd           press.h:317-320 */
	PM_SVB12_ZD         = 1,  /* svb16-zd               press.h:348-352 */
	PM_SVB_ZD           = 2,  /* svb-zd (svb32)         press.h:324-328 */
	PM_ZSTD_SVB_ZD      = 3,  /* VBZ                    press.h:380-384 */
	PM_ZSTD_SVB12_ZD    = 4,  /* zstd-svb16-zd          press.h:404-408 */
	PM_VBE21_ZD         = 5,  /* press.h:498-502 */
	PM_VBBE21_ZD        = 6,  /* press.h:506-510 */
	PM_VBSBE21_ZD       = 7,  /* press.h:514-518 */
	PM_VBSSE21_ZD       = 8,  /* press.h:522-526 */
	PM_SHUFF_VBE21_ZD   = 9,  /* press.h:634-638 */
	PM_SHUFF_VBBE21_ZD  = 10, /* press.h:642-646 */
	PM_SHUFF_VBSBE21_ZD = 11, /* press.h:650-654 */
	PM_SHUFF_VBSSE21_ZD = 12, /* press.h:658-662 */
	PM_HASGAM_ZDQ       = 13, /* ex-zd                  press.h:960-964 */
	PM_ZSTD_HASGAM_ZDQ  = 14, /* zstd over ex-zd        press.c:8554 */
	PM_SLOW5_SVB_ZD     = 15, /* BLOW5 signal codec "svb-zd": slow5lib slow5_press.c:1054,1110 */
	PM_RC_VBE21_ZD      = 16, /* vbe21 + order-0 range coder (TurboRC rcsenc)  press.h:712-716 */
	PM_RCC_VBE21_ZD     = 17, /* vbe21 + order-1 range coder (TurboRC rccsenc) press.h (press.c:5510-5580) */
	PM_RCCM_VBBE21_ZD   = 18, /* vbbe21 + order 1-0 context mixing with SSE (TurboRC rcmsenc) press.c:6901-7000 */
	PM_NMETHODS         = 19
};

#endif
