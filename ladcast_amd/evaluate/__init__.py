from .utils import (ensemble_scores, get_acc, get_crps, get_lat_weights_from_lat_tensor, get_normalized_lat_weights_based_on_cos,
                    pointwise_crps_skill, pointwise_crps_spread)
