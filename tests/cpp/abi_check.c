/* abi_check.c -- include/gsr.h compiled as plain C (-std=c99 -pedantic) and linked against libgsr_hip.so: the header is a
 * C header, every entry point links with C linkage, and the host-only calls behave.  No device work (runs without a GPU). */
#include <stdio.h>
#include <string.h>

#include "gsr.h"

/* one reference to every entry point of the header, so that a missing export fails at link time */
typedef void (*any_fn)(void);
static const any_fn entry_points[] = {
	(any_fn)gsr_last_error, (any_fn)gsr_version, (any_fn)gsr_thread_release, (any_fn)gsr_geometry_bytes,
	(any_fn)gsr_image_bytes, (any_fn)gsr_binning_bytes, (any_fn)gsr_backward_scratch_bytes,
	(any_fn)gsr_geometry_layout_of, (any_fn)gsr_image_layout_of, (any_fn)gsr_binning_layout_of,
	(any_fn)gsr_forward_preprocess, (any_fn)gsr_forward_render, (any_fn)gsr_backward, (any_fn)gsr_backward_blend,
	(any_fn)gsr_backward_gaussians, (any_fn)gsr_sh_grad_from_views, (any_fn)gsr_loss_scratch_bytes,
	(any_fn)gsr_l1_ssim_loss, (any_fn)gsr_forward_preprocess_leaf, (any_fn)gsr_backward_leaf, (any_fn)gsr_adam_step,
	(any_fn)gsr_knn_scratch_bytes, (any_fn)gsr_knn_mean_dist2, (any_fn)gsr_mark_visible, (any_fn)gsr_get_higher_msb,
	(any_fn)gsr_profile_begin, (any_fn)gsr_profile_begin_only, (any_fn)gsr_profile_end,
};

int main(void)
{
	gsr_geometry_layout gl;
	gsr_backward_args args;
	int64_t R = 0;
	size_t k, n = sizeof entry_points / sizeof entry_points[0];
	for (k = 0; k < n; k++)
		if (!entry_points[k]) return 1;
	if (!strstr(gsr_version(), "gfx950")) return 2;
	if (gsr_get_higher_msb(8432u) != 14u || gsr_get_higher_msb(32400u) != 15u) return 3;   /* rasterizer_impl.cu:37-52 at 1080p / 4K */
	if (gsr_geometry_layout_of(1000, &gl) != GSR_OK || gl.total != gsr_geometry_bytes(1000) || gl.total < 48u * 1000u) return 4;
	if (gsr_image_bytes(1980, 1080) < (size_t)8 * 1980 * 1080) return 5;
	if (gsr_backward_scratch_bytes(10, 100) < (size_t)48 * 100) return 6;
	/* argument validation happens before any device work */
	if (gsr_forward_preprocess(-1, 0, 0, 16, 16, NULL, NULL, NULL, NULL, NULL, 1.0f, NULL, NULL, NULL, NULL, NULL, 1.0f, 1.0f, 0, NULL, NULL,
	                           &R, NULL, 0) != GSR_ERR_INVALID_ARGUMENT || !strstr(gsr_last_error(), "bad")) return 7;
	memset(&args, 0, sizeof args);
	args.P = 4; args.width = 16; args.height = 16;
	if (gsr_backward_blend(&args) != GSR_ERR_INVALID_ARGUMENT) return 8;
	if (gsr_thread_release() != GSR_OK) return 9;   /* nothing was created: a no-op */
	printf("abi_check ok: %s, %u entry points, sizeof(gsr_backward_args) = %u\n", gsr_version(), (unsigned)n, (unsigned)sizeof args);
	return 0;
}
