"""Ground-truth label conversions used by the host feeder
(reference: helpers/ground_truth_conversion_utils.py:3-88).  `convert_IDs_to_one_hot` defines the
label layout the loss consumes; the library itself ships uint8 class ids to the GPU."""
import numpy as np


def convert_IDs_to_IDs(input_array, id_map_array):
    """LUT gather: `id_map_array[input_array]` (:3-24)."""
    return np.asarray(id_map_array)[input_array]


def convert_IDs_to_IDs_partial(image, id_map_dict):
    """Replace only the ids present in `id_map_dict` (:26-52; the reference version raises
    NameError because it iterates an undefined `id_map`)."""
    out = np.copy(image)
    for key, value in id_map_dict.items():
        out[image == key] = value
    return out


def convert_between_IDs_and_colors(image, color_map_dict, gt_dtype=np.uint8):
    """3-channel colours -> ids, or ids -> colours, by dictionary lookup (:54-65)."""
    if np.squeeze(image).ndim == 3:
        out = np.zeros(image.shape[:2], dtype=gt_dtype)
        for key, value in color_map_dict.items():
            out[np.all(image == key, axis=2)] = value
        return out
    out = np.zeros(image.shape[:2] + (3,), dtype=np.uint8)
    for key, value in color_map_dict.items():
        out[image == key] = value
    return out


def convert_IDs_to_colors(image, color_map_array):
    """`color_map_array[image]` (:67-78)."""
    return np.asarray(color_map_array)[image]


def convert_one_hot_to_IDs(one_hot):
    """argmax over the last axis, squeezed (:80-82)."""
    return np.squeeze(np.argmax(one_hot, axis=-1))


def convert_IDs_to_one_hot(image, num_classes):
    """`np.eye(num_classes, dtype=bool)[image]` (:84-88)."""
    return np.eye(num_classes, dtype=bool)[image]
