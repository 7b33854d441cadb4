"""ransac.jl_amd -- MI355X-native (gfx950 HIP) hot path of Efficient RANSAC behind the API of
cserteGT3/RANSAC.jl: FittedShape / RANSACCloud / ransac().  Import as `ransac_jl_amd`."""
from . import _lib, synth
from ._lib import (CONE, CYLINDER, PLANE, SPHERE, SCORE_F64, SCORE_INT64_WRAP, RansacHipError, lib)
from .api import (DEFAULT_PARAMETERS, DEFAULT_SHAPE_DICT, ConfidenceInterval, E, ExtractedShape, FittedCone,
                  IterationCandidates, deleteat, findhighestscore, forcefitshapes, push2candidatesandlevels, recordscore,
                  removeinvalidshapes, scorecandidates, setfloattype, findAABB, smallestdistance,
                  FittedCylinder, FittedPlane, FittedShape, FittedSphere, RANSACCloud, bitmapparameters,
                  defaultcommonparameters, defaultiterationparameters, defaultparameters,
                  defaultshapeparameters, estimatescore, fit, invalidate_indexes, largestconncomp,
                  notsoconfident, params_to_c, prob, ransac, ransacparameters, MpGroup, refit, refit_lsq, score_batch,
                  scorecandidate, select_enabled, sample_sets, fit_sets, shape_f32, shape_from_c, strt, buildoctree, octreedepth, findleaf,
                  getnthcell, iswithinrectangle, cell_enabled_points, OctreeCell, set_option, get_option, option)

from .io import exportJSON, readconfig, toDict

__all__ = [n for n in dir() if not n.startswith("_")]
