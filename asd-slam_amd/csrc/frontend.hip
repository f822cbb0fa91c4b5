#include "ctx.h"
int frontend_alloc(asd_ctx*) { return ASD_OK; }
void frontend_free(asd_ctx*) {}
