"""MI355X-native versions of the InfiniCube utilities that produce the hot path's inputs."""
