"""Host-side mirror of the one hot-path member of the reference's
`helper_tool.DataProcessing` (ffb6d/models/RandLA/helper_tool.py:160-170)."""
import numpy as np

from . import nearest_neighbors


class DataProcessing:
    @staticmethod
    def knn_search(support_pts, query_pts, k):
        """
        :param support_pts: points you have, B*N1*3
        :param query_pts: points you want to know the neighbour index, B*N2*3
        :param k: Number of neighbours in knn search
        :return: neighbor_idx: neighboring points indexes, B*N2*k  (int32)

        numpy arrays run through `cpp_knn_batch_omp`; torch GPU tensors stay on the device
        (int32 tensor back), which is the path the on-device pyramid builder uses.
        """
        if isinstance(support_pts, np.ndarray):
            idx = nearest_neighbors.knn_batch(support_pts, query_pts, k, omp=True)
            return idx.astype(np.int32)
        import torch
        return nearest_neighbors.knn_batch_device(support_pts, query_pts, k, dtype=torch.int32)
