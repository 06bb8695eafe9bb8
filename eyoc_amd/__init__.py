"""eyoc_amd - MI355X-native implementation of EYOC's registration hot path.

Public names follow the reference (liuQuan98/EYOC): ``SparseTensor``, ``load_model`` /
``ResUNetBN2C``, ``find_nn_gpu`` / ``pdist`` / ``find_corr``, ``est_quad_linear_robust`` /
``pose_estimation``, ``rigid_transform_3d``, ``Matcher``; plus the two aliases named by the project
brief, ``find_correspondences`` and ``estimate_transform``.  All compute happens in
``eyoc_amd/lib/libeyoc_hip.so`` (hand-written HIP for gfx950) through the C ABI of
``include/eyoc_hip.h``; there is no CPU fallback.
"""
from ._lib import EyocError, LIB_PATH  # noqa: F401
from .sparse_tensor import SparseTensor, CoordinateManager  # noqa: F401
from .model import (load_model, ResUNet2, ResUNetBN2, ResUNetBN2B, ResUNetBN2C, ResUNetBN2D,  # noqa: F401
                    ResUNetBN2E, ResUNetFatBN, MODELS)
from .eval import find_nn_gpu, pdist, find_corr, find_correspondences, random_sample, knn1_segmented  # noqa: F401
from .transform_estimation import (est_quad_linear_robust, estimate_transform, pose_estimation,  # noqa: F401
                                   rigid_transform_3d, transform, integrate_trans)
from .registration import (Matcher, registration_ransac_based_on_feature_matching,  # noqa: F401
                           ransac_from_correspondences, ransac_batched_from_correspondences, RegistrationResult)
from .metrics import registration_errors, apply_transform, evaluate_nn_dist  # noqa: F401
from .voxelize import sparse_quantize, voxelize, extract_features  # noqa: F401
from .labels import (knn2_segmented, lowe_topk, spherical_filter, similarity_filter, load_dist_sim_map,  # noqa: F401
                     match_and_filter_corr, correspondences_under_pose)
from .autograd import sparse_conv, contrastive_hardest_negative_loss  # noqa: F401,E402
