// placeholder until the fused kernel lands
#include "mtk_common.h"
int mtk_pack_fused(mtk_ctx* c, hipStream_t) { c->fused_ok = false; return MTK_OK; }
void mtk_free_fused(mtk_ctx* c) { c->fused_ok = false; }
int mtk_fused_forward(mtk_ctx*, int64_t, const float*, int64_t, int64_t, const int64_t*, int, float*, int64_t, int64_t, hipStream_t) { MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused kernel not built"); }
int mtk_fused_tf_score(mtk_ctx*, int64_t, int, const float*, int64_t, const float*, int, const int32_t*, int, float*, hipStream_t) { MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused kernel not built"); }
