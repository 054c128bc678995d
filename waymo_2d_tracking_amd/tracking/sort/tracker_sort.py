#
# Multi class SORT tracker - mirrors /root/reference/tracking/sort/tracker_sort.py
#
import numpy as np

from .sort import Sort


class MultiClassTrackerSort(object):

    def __init__(self, max_age=1, min_hits=0):
        """tracker_sort.py:12-20: one GPU-resident Sort per class, created on first sight."""
        self.max_age = max_age
        self.min_hits = min_hits
        self.trackers = {}

    def track(self, detected_objects, iou_thresholds):
        """tracker_sort.py:22-51
        :param detected_objects: [[x1, y1, x2, y2, confidence, class_name], ...]
        :return: {class_name: ndarray (K,6) [x1, y1, x2, y2, object_id, confidence]}
        """
        class2detections = {}
        for detected_object in detected_objects:
            class_name = detected_object[5]
            if class_name not in self.trackers:
                self.trackers[class_name] = Sort(max_age=self.max_age, min_hits=self.min_hits)
            class2detections.setdefault(class_name, []).append(detected_object[:5])

        all_tracked_objects = {}
        for class_name, class_tracker in self.trackers.items():      # first-seen order
            dets = np.array(class2detections.get(class_name, []), dtype=np.float32)
            all_tracked_objects[class_name] = class_tracker.update(dets, iou_threshold=iou_thresholds[class_name - 1])
        return all_tracked_objects
