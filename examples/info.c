/* examples/info.c -- the C-ABI from plain C99: headers and geometry of a JPEG file, on the host (no GPU), then,
 * where a GPU is present, the file decoded through the drop-in for the reference's process_jpeg_file().
 *
 *   gcc -std=c99 -pedantic -Wall -Iinclude examples/info.c -Ljpeg_amd -ljpeg_amd -Wl,-rpath,$PWD/jpeg_amd -o /tmp/info
 *   /tmp/info tests/golden/d_pil_422_jfif.jpg [out.ppm]
 */
#include <stdio.h>
#include <stdlib.h>
#include "jpeg_amd.h"

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: %s in.jpg [out.ppm]\n", argv[0]);
		return 2;
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f) {
		return RET_FAILURE_FILE_OPEN & 0xff;
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *buf = malloc((size_t)n);
	if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) {
		return 1;
	}
	fclose(f);

	jpeg_hip_frame_info fi;
	int err = jpeg_hip_parse_info(buf, (size_t)n, 0, &fi);
	if (err) {
		printf("%s: RET 0x%x\n", argv[1], (unsigned)err);
		return 1;
	}
	printf("%s: %dx%d P=%d Nf=%d", jpeg_hip_version(), fi.X, fi.Y, fi.P, fi.Nf);
	for (int k = 0; k < fi.Nf; ++k) {
		printf(" [id %d %dx%d Tq %d]", fi.id[k], fi.H[k], fi.V[k], fi.Tq[k]);
	}
	printf(" MCUs %dx%d frames %ld\n", fi.m_x, fi.m_y, (long)jpeg_hip_mjpeg_split(buf, (size_t)n, NULL, NULL, 0));
	free(buf);

	if (argc > 2) {
		if (jpeg_hip_device_count() <= 0) {
			printf("no HIP device: not decoding (the library has no CPU path)\n");
			return 0;
		}
		err = jpeg_hip_process_jpeg_file(argv[1], argv[2]);            /* decoder.c:687 */
		printf("decoded to %s: RET 0x%x\n", argv[2], (unsigned)err);
		return err ? 1 : 0;
	}
	return 0;
}
